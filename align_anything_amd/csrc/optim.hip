// Flat-buffer optimizer kernels for gfx950 (HBM-bound, 28 B/param/step).
// Replaces DeepSpeed FusedAdam (adam_w_mode, bias_correction) + global-L2 gradient clipping used at
// align_anything/trainers/base/supervised_trainer.py:245-249 with configs/deepspeed/*.json
// "gradient_clipping".  MI355X-first layout: all parameters of one group live in ONE flat bf16
// buffer with matching flat fp32 master/m/v buffers, so a step is a handful of streaming launches
// instead of a multi-tensor pointer table.
#include "aa_common.h"

// sum of squares of a flat gradient buffer, accumulated into *out (fp32).  DETERMINISTIC: every block writes one
// partial into ws[blockIdx.x], a single-block kernel adds them in a fixed tree -- no float atomics, so the clip
// coefficient (and with it every weight) is bit-identical on all data-parallel ranks and from run to run.
// scale is applied before squaring (e.g. 1/world after a SUM all-reduce).
template <typename TG>
__global__ __launch_bounds__(256) void sumsq_kernel(const TG* __restrict__ g, long n, float scale,
                                                    float* __restrict__ ws) {
    __shared__ float red[8];
    float acc = 0.f;
    if constexpr (sizeof(TG) == 2) {
        const long n8 = n >> 3;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
            u16x8 v = *reinterpret_cast<const u16x8*>(g + i * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float f = bf2f(v[j]) * scale; acc += f * f; }
        }
        for (long i = (n8 << 3) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
            const float f = bf2f(g[i]) * scale; acc += f * f;
        }
    } else {
        const long n4 = n >> 2;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
            f32x4 v = *reinterpret_cast<const f32x4*>(g + i * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float f = v[j] * scale; acc += f * f; }
        }
        for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
            const float f = g[i] * scale; acc += f * f;
        }
    }
    acc = block_sum<256>(acc, red);
    if (threadIdx.x == 0) ws[blockIdx.x] = acc;
}

__global__ __launch_bounds__(256) void sumsq_finish_kernel(const float* __restrict__ ws, int nparts,
                                                           float* __restrict__ out) {
    __shared__ float red[8];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += ws[i];
    acc = block_sum<256>(acc, red);
    if (threadIdx.x == 0) *out += acc;
}

extern "C" int aa_grad_sumsq(const void* g, int g_dtype, long n, float scale, float* out_accum,
                             float* ws, void* stream) {
    AA_REQUIRE(g_dtype == 0 || g_dtype == 1, "aa_grad_sumsq: dtype must be 0 (bf16) or 1 (f32)");
    AA_REQUIRE(ws != nullptr, "aa_grad_sumsq: ws (AA_SUMSQ_WS floats of scratch) is required");
    if (n == 0) return AA_OK;
    const long work = n / 8 / 256 + 1;
    const int grid = (int)(work < 2048 ? work : 2048);
    if (g_dtype == 0)
        hipLaunchKernelGGL(sumsq_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)g, n, scale, ws);
    else
        hipLaunchKernelGGL(sumsq_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const float*)g, n, scale, ws);
    hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ws, grid, out_accum);
    AA_CHECK_LAUNCH("aa_grad_sumsq");
    return AA_OK;
}

// Reduce step of the DIRECT gradient exchange (engine.GradReducer mode 'direct': all-to-all of the w chunks over the w - 1 point-to-point xGMI links,
// this sum, all-gather): out[i] = sum over r of in[r][i] for i < c, the w chunks summed in rank order in fp32 and rounded ONCE to the gradient dtype --
// every rank computes its own chunk the same way, so after the all-gather the replicas hold identical bits (and a bf16 bucket is rounded once instead
// of w - 1 times along a ring).  HBM-bound: (w + 1) x c elements.  SURVEY.md section 8(e): "7-link-parallel reduce-scatter / all-gather".
template <typename T>
__global__ __launch_bounds__(256) void chunk_sum_kernel(const T* __restrict__ in, T* __restrict__ out, long c, int w) {
    const long c8 = c >> 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < c8; i += (long)gridDim.x * 256) {
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int r = 0; r < w; ++r) {
            if constexpr (sizeof(T) == 2) {
                const u16x8 v = *reinterpret_cast<const u16x8*>(in + (long)r * c + i * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += bf2f(v[e]);
            } else {
                const f32x4 a = *reinterpret_cast<const f32x4*>(in + (long)r * c + i * 8), b = *reinterpret_cast<const f32x4*>(in + (long)r * c + i * 8 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[e] += a[e]; acc[4 + e] += b[e]; }
            }
        }
        if constexpr (sizeof(T) == 2) {
            u16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f2bf(acc[e]);
            *reinterpret_cast<u16x8*>(out + i * 8) = o;
        } else {
            *reinterpret_cast<f32x4*>(out + i * 8) = f32x4{acc[0], acc[1], acc[2], acc[3]};
            *reinterpret_cast<f32x4*>(out + i * 8 + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
        }
    }
}
extern "C" int aa_chunk_sum(const void* in, void* out, int dtype, long chunk, int world, void* stream) {
    AA_REQUIRE(dtype == 0 || dtype == 1, "aa_chunk_sum: dtype must be 0 (bf16) or 1 (f32)");
    AA_REQUIRE(world >= 1 && chunk >= 0 && chunk % 8 == 0, "aa_chunk_sum: world %d, chunk %ld (a multiple of 8 elements)", world, chunk);
    AA_REQUIRE((((uintptr_t)in | (uintptr_t)out) & 15) == 0, "aa_chunk_sum: buffers must be 16-byte aligned");
    if (chunk == 0) return AA_OK;
    const long work = chunk / 8 / 256 + 1;
    const int grid = (int)(work < 4096 ? work : 4096);
    if (dtype == 0) hipLaunchKernelGGL(chunk_sum_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, (bf16_t*)out, chunk, world);
    else hipLaunchKernelGGL(chunk_sum_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)in, (float*)out, chunk, world);
    AA_CHECK_LAUNCH("aa_chunk_sum");
    return AA_OK;
}

// clip_coef = min(1, max_norm / (sqrt(sumsq) + 1e-6)) ; norm_out = sqrt(sumsq)  (device-side, no host sync)
// sumsq == -inf is the "skip this update" sentinel (the expert-parallel capacity overflow, all-reduced into the buffer by the engine:
// expert_parallel.py): coef = norm = -1, and the AdamW kernels below return without touching weights or moments on a negative coefficient.
// A squared norm can be +inf or NaN (diverged gradients: those propagate as before) but never negative.
__global__ void clip_coef_kernel(const float* __restrict__ sumsq, float max_norm,
                                 float* __restrict__ coef, float* __restrict__ norm_out) {
    if (*sumsq == -INFINITY) {
        if (norm_out) *norm_out = -1.f;
        *coef = -1.f;
        return;
    }
    const float nrm = sqrtf(*sumsq);
    if (norm_out) *norm_out = nrm;
    float c = 1.f;
    if (max_norm > 0.f) c = fminf(1.f, max_norm / (nrm + 1e-6f));
    *coef = c;
}
extern "C" int aa_clip_coef(const float* sumsq, float max_norm, float* coef_out, float* norm_out,
                            void* stream) {
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sumsq, max_norm,
                       coef_out, norm_out);
    AA_CHECK_LAUNCH("aa_clip_coef");
    return AA_OK;
}

// AdamW on flat buffers.  g' = g * gscale * (*clip_coef)
//   m = b1 m + (1-b1) g' ; v = b2 v + (1-b2) g'^2
//   p = p - lr * ( (m/bc1) / (sqrt(v/bc2) + eps) + wd * p )       (DeepSpeed FusedAdam ADAM_MODE_1)
// master/m/v fp32, p16 = bf16 shadow written from the updated master (NULL in the fp32 parity mode, where the
// model computes directly on the fp32 masters).
template <typename TG>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ master, float* __restrict__ m,
                                                    float* __restrict__ v, bf16_t* __restrict__ p16,
                                                    const TG* __restrict__ g, long n, float lr,
                                                    float b1, float b2, float eps, float wd,
                                                    float bc1, float bc2, float gscale,
                                                    const float* __restrict__ clip_coef) {
    const float gs = gscale * (clip_coef ? *clip_coef : 1.f);
    if (gs < 0.f) return;          // clip_coef_kernel's skip sentinel: this step's update is dropped on every rank
    const float inv_bc1 = 1.f / bc1, inv_bc2 = 1.f / bc2;
    const long n4 = n >> 2;
    // Round 5: every operand is touched exactly once per step (189 GB at 7B: nothing here is worth a cache line) -> nontemporal loads and stores, and TWO
    // independent 16-element groups per thread and iteration (all eight loads requested before the first use).  Same expressions per element (the compiler's fma
    // contraction may differ by an ulp from round 4's loop shape).  Same box, alternating: 7.79 / 7.83 -> 7.40 / 7.68 ms per 1.6 G elements (5.78 -> 5.98 TB/s);
    // the step at 1 pair 218.8 -> 217.6 ms, at 4 pairs inside the noise (the update is fully exposed either way: profiles/r05_adam_window.txt).
    auto update4 = [&](f32x4& pw, f32x4& mm, f32x4& vv, const float (&gg)[4], u16x4& o) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            mm[j] = b1 * mm[j] + (1.f - b1) * gg[j];
            vv[j] = b2 * vv[j] + (1.f - b2) * gg[j] * gg[j];
            const float denom = sqrtf(vv[j] * inv_bc2) + eps;
            const float upd = (mm[j] * inv_bc1) / denom + wd * pw[j];
            pw[j] = pw[j] - lr * upd;
            o[j] = f2bf(pw[j]);
        }
    };
    const long stride = (long)gridDim.x * 256;
    for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += 2 * stride) {
        const long i1 = i0 + stride;
        const bool two = i1 < n4;
        f32x4 pw[2], mm[2], vv[2];
        float gg[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) break;
            const long i = u ? i1 : i0;
            pw[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(master + i * 4));
            mm[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(m + i * 4));
            vv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(v + i * 4));
            if constexpr (sizeof(TG) == 2) {
                const u16x4 gr = __builtin_nontemporal_load(reinterpret_cast<const u16x4*>(g + i * 4));
#pragma unroll
                for (int j = 0; j < 4; ++j) gg[u][j] = bf2f(gr[j]) * gs;
            } else {
                const f32x4 gr = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g + i * 4));
#pragma unroll
                for (int j = 0; j < 4; ++j) gg[u][j] = gr[j] * gs;
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) break;
            const long i = u ? i1 : i0;
            u16x4 o;
            update4(pw[u], mm[u], vv[u], gg[u], o);
            __builtin_nontemporal_store(pw[u], reinterpret_cast<f32x4*>(master + i * 4));
            __builtin_nontemporal_store(mm[u], reinterpret_cast<f32x4*>(m + i * 4));
            __builtin_nontemporal_store(vv[u], reinterpret_cast<f32x4*>(v + i * 4));
            if (p16) __builtin_nontemporal_store(o, reinterpret_cast<u16x4*>(p16 + i * 4));
        }
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float gq;
        if constexpr (sizeof(TG) == 2) gq = bf2f(g[i]) * gs; else gq = g[i] * gs;
        const float mq = b1 * m[i] + (1.f - b1) * gq;
        const float vq = b2 * v[i] + (1.f - b2) * gq * gq;
        const float denom = sqrtf(vq * inv_bc2) + eps;
        const float pq = master[i] - lr * ((mq * inv_bc1) / denom + wd * master[i]);
        master[i] = pq; m[i] = mq; v[i] = vq;
        if (p16) p16[i] = f2bf(pq);
    }
}

// "Thin" variant for running UNDER the GEMMs of the next step (NativeEngine.step launches the optimizer on a side stream while the
// main stream already runs the reference forward): the 256x256 GEMM tile holds 2 waves x 248 VGPRs per SIMD lane, which leaves
// 16 of the 512 -- a wave that needs at most 16 VGPRs (and no LDS) can be co-resident with them, so this HBM-bound update
// overlaps the MFMA-bound GEMMs instead of waiting for a CU to drain.  Two elements per lane, everything else in SGPRs.
template <typename TG>
__global__ __launch_bounds__(256)
void adamw_thin_kernel(float* __restrict__ master, float* __restrict__ m, float* __restrict__ v, bf16_t* __restrict__ p16,
                       const TG* __restrict__ g, uint32_t n, float lr, float b1, float b2, float eps, float wd, float inv_bc1,
                       float inv_bc2, float gscale, const float* __restrict__ clip_coef) {
    // buffer addressing: the five base addresses live in SGPR descriptors, each access costs ONE VGPR byte offset
    // (the host chunks the flat buffer so offsets fit 32 bits); one element per lane.
    const float gs = gscale * (clip_coef ? *clip_coef : 1.f);
    if (gs < 0.f) return;          // skip sentinel (see clip_coef_kernel)
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(master, 0, n * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(m, 0, n * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(v, 0, n * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<TG*>(g), 0, n * (uint32_t)sizeof(TG), 0x00020000);
    const __amdgpu_buffer_rsrc_t r16 = __builtin_amdgcn_make_buffer_rsrc(p16 ? p16 : (bf16_t*)master, 0, p16 ? n * 2u : 0u, 0x00020000);
    const uint32_t stride = gridDim.x * 256u;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += stride) {
        const uint32_t o4 = i * 4u;
        float gg;
        if constexpr (sizeof(TG) == 2) gg = bf2f((bf16_t)__builtin_amdgcn_raw_buffer_load_b16(rg, i * 2u, 0, 0)) * gs;
        else gg = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, o4, 0, 0)) * gs;
        const float mm = b1 * __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rm, o4, 0, 0)) + (1.f - b1) * gg;
        const float vv = b2 * __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, o4, 0, 0)) + (1.f - b2) * gg * gg;
        float pw = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, o4, 0, 0));
        // hardware sqrt / rcp (1 ulp): keeps the live range under 16 VGPRs; the update is far below bf16 resolution either way
        pw -= lr * ((mm * inv_bc1) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(vv * inv_bc2) + eps) + wd * pw);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, mm), rm, o4, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, vv), rv, o4, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, pw), rp, o4, 0, 0);
        if (p16) __builtin_amdgcn_raw_buffer_store_b16(f2bf(pw), r16, i * 2u, 0, 0);
    }
}

extern "C" int aa_adamw_set_thin(int on) { aa_ctx_cur()->adam_thin = on ? 1 : 0; return AA_OK; }

extern "C" int aa_adamw_flat(float* master, float* m, float* v, void* p16, const void* g, int g_dtype,
                             long n, float lr, float beta1, float beta2, float eps, float weight_decay,
                             int step, float gscale, const float* clip_coef, void* stream) {
    AA_REQUIRE(g_dtype == 0 || g_dtype == 1, "aa_adamw_flat: dtype must be 0 (bf16) or 1 (f32)");
    AA_REQUIRE(step >= 1, "aa_adamw_flat: step must be >= 1 (got %d)", step);
    if (n == 0) return AA_OK;
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2 = (float)(1.0 - pow((double)beta2, (double)step));
    const long work = n / 4 / 256 + 1;
    const int grid = (int)(work < 4096 ? work : 4096);
    hipStream_t st = (hipStream_t)stream;
    if (aa_ctx_cur()->adam_thin) {
        const long CH = 1L << 28;          // elements per launch: 32-bit offsets inside the kernel
        for (long o = 0; o < n; o += CH) {
            const uint32_t cn = (uint32_t)((n - o) < CH ? (n - o) : CH);
            const int tg = (int)((cn + 255u) / 256u < 16384u ? (cn + 255u) / 256u : 16384u);
            bf16_t* p16o = p16 ? (bf16_t*)p16 + o : nullptr;
            if (g_dtype == 0)
                hipLaunchKernelGGL(adamw_thin_kernel<bf16_t>, dim3(tg), dim3(256), 0, st, master + o, m + o, v + o, p16o, (const bf16_t*)g + o, cn,
                                   lr, beta1, beta2, eps, weight_decay, 1.f / bc1, 1.f / bc2, gscale, clip_coef);
            else
                hipLaunchKernelGGL(adamw_thin_kernel<float>, dim3(tg), dim3(256), 0, st, master + o, m + o, v + o, p16o, (const float*)g + o, cn,
                                   lr, beta1, beta2, eps, weight_decay, 1.f / bc1, 1.f / bc2, gscale, clip_coef);
        }
        AA_CHECK_LAUNCH("aa_adamw_flat");
        return AA_OK;
    }
    if (g_dtype == 0)
        hipLaunchKernelGGL(adamw_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, master, m, v,
                           (bf16_t*)p16, (const bf16_t*)g, n, lr, beta1, beta2, eps, weight_decay, bc1,
                           bc2, gscale, clip_coef);
    else
        hipLaunchKernelGGL(adamw_kernel<float>, dim3(grid), dim3(256), 0, st, master, m, v,
                           (bf16_t*)p16, (const float*)g, n, lr, beta1, beta2, eps, weight_decay, bc1,
                           bc2, gscale, clip_coef);
    AA_CHECK_LAUNCH("aa_adamw_flat");
    return AA_OK;
}
