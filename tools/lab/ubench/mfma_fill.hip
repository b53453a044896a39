// In-wave overlap of v_mfma_f32_32x32x16_bf16 with VALU fillers on gfx950, one wave per SIMD (lab; run through gpurun):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_fill tools/lab/ubench/mfma_fill.hip && /tmp/mfma_fill
// Each variant: a loop of 16 MFMAs per trip over 8 independent accumulators, F fillers after every MFMA; s_memtime around 64 trips; cycles per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define MFMA_A(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(B))
#define MFMA_V(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))
#define MFMA_VQ(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "a"(B))
#define FENCE __builtin_amdgcn_sched_barrier(0)

// KIND: 0 = acc in AGPR (8 independent), 1 = acc in VGPR (8 independent), 2 = acc in VGPR, 2 chains only, B operand in AGPR (the score chains)
// FILL: 0 = v_fma_f32, 1 = v_exp_f32, 2 = mix exp/add/cvt like the E stream, 3 = ds_read_b128
template <int KIND, int F, int FILL>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned* cyc, int trips) {
    extern __shared__ char smem[];
    f32x16 acc[8];
    bf16x8 a[4], b[4];
    float x[8];
    const int lane = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = lane * 0.001f + i; for (int r = 0; r < 16; ++r) acc[i][r] = 0.f; }
#pragma unroll
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) { a[i][e] = (__bf16)(0.001f * (lane + e + i)); b[i][e] = (__bf16)(0.002f * (lane - e + i)); }
    const int laddr = (int)(uintptr_t)smem + (lane & 63) * 16;
    FENCE;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    FENCE;
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if constexpr (KIND == 0) MFMA_A(acc[m & 7], a[m & 3], b[(m >> 1) & 3]);
            else if constexpr (KIND == 1) MFMA_V(acc[m & 7], a[m & 3], b[(m >> 1) & 3]);
            else MFMA_VQ(acc[m & 1], a[m & 3], b[(m >> 1) & 3]);
#pragma unroll
            for (int f = 0; f < F; ++f) {
                if constexpr (FILL == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[f & 7]) : "v"(x[(f + 3) & 7]));
                else if constexpr (FILL == 1) asm volatile("v_exp_f32 %0, %1" : "=v"(x[f & 7]) : "v"(x[(f + 3) & 7]));
                else if constexpr (FILL == 2) {
                    if (f % 3 == 0) asm volatile("v_exp_f32 %0, %1" : "=v"(x[f & 7]) : "v"(x[(f + 3) & 7]));
                    else if (f % 3 == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[(f + 4) & 7]) : "v"(x[(f + 1) & 7]));
                    else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x[(f + 5) & 7]) : "v"(x[(f + 2) & 7]), "v"(x[f & 7]));
                } else {
                    bf16x8 r;
                    asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(laddr));
                    asm volatile("" :: "v"(r));
                }
            }
            FENCE;
        }
        if constexpr (FILL == 3) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    FENCE;
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { s += x[i]; for (int r = 0; r < 16; ++r) s += acc[i][r]; }
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = (unsigned)(t1 - t0);
}

template <int KIND, int F, int FILL>
void run(const char* name, float* out, unsigned* cyc) {
    const int trips = 64, blocks = 256;
    hipFuncSetAttribute((const void*)k<KIND, F, FILL>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<KIND, F, FILL>), dim3(blocks), dim3(256), 96 * 1024, 0, out, cyc, trips);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, F, FILL>), dim3(blocks), dim3(256), 96 * 1024, 0, out, cyc, trips);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned> h(blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 4, hipMemcpyDeviceToHost);
    double sum = 0; for (unsigned v : h) sum += v;
    const double per = sum / h.size() / (trips * 16);
    printf("%-34s F=%d  %6.1f ticks/MFMA   (kernel %.1f us -> %.2f GHz-equivalent if ticks are cycles)\n", name, F, per, ms * 1e3, sum / h.size() / (ms * 1e3) / 1e3);
}

int main() {
    float* out; unsigned* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 4 * 4);
#define RUNF(K, FILL, NAME) run<K, 0, FILL>(NAME, out, cyc); run<K, 2, FILL>(NAME, out, cyc); run<K, 4, FILL>(NAME, out, cyc); run<K, 5, FILL>(NAME, out, cyc); run<K, 6, FILL>(NAME, out, cyc); run<K, 8, FILL>(NAME, out, cyc);
    RUNF(0, 0, "acc AGPR x8, fill v_fma")
    RUNF(0, 1, "acc AGPR x8, fill v_exp")
    RUNF(0, 2, "acc AGPR x8, fill exp/add/cvt")
    RUNF(1, 0, "acc VGPR x8, fill v_fma")
    RUNF(2, 0, "acc VGPR 2 chains B=AGPR, v_fma")
    RUNF(2, 2, "acc VGPR 2 chains B=AGPR, e/a/c")
    run<0, 1, 3>("acc AGPR x8, fill ds_read_b128", out, cyc); run<0, 2, 3>("acc AGPR x8, fill ds_read_b128", out, cyc);
    return 0;
}
