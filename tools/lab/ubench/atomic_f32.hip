// Lab microbenchmark (round 6): throughput of fire-and-forget fp32 global atomics in the pattern a single-pass attention backward would use for dQ.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/lab/ubench/atomic_f32.hip -o gpurun_out/atomic_f32 && gpurun_out/atomic_f32
// Every workgroup (4 waves) owns a (head, key block) pair and walks the head's query tiles; per tile each wave adds a [64 q][32 d] fp32 block (its d slice
// of the [64][128] dQ tile) = 32 wave-wide atomics.  Modes: 0 = 32x32 accumulator layout (2 rows x 128 B per instruction), 1 = 16x16 layout (4 rows x 64 B),
// 2 = one contiguous 256 B row piece per instruction, 3 = plain stores in layout 0 (the bandwidth the same bytes cost without the read-modify-write).
// XCD-local = all workgroups of a head on one XCD (block b -> XCD b % 8).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE, bool XCD_LOCAL>
__global__ __launch_bounds__(256) void atomic_kernel(float* buf, int heads, int T, int nkb, float v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x;
    int head, kb;
    if (XCD_LOCAL) { head = (b & 7) + 8 * ((b >> 3) / nkb); kb = (b >> 3) % nkb; }
    else { head = b % heads; kb = b / heads; }
    float* hb = buf + (long)head * T * 128;
    const int ntq = T / 64;
    for (int qt = kb * (ntq / nkb) % ntq, it = 0; it < ntq - kb * (ntq / nkb); ++it, qt = (qt + 1) % ntq) {      // causal: key block kb sees the query tiles at and after it
        float* tile = hb + (long)qt * 64 * 128 + wave * 32;
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            int row, col;
            if (MODE == 0 || MODE == 3) { row = (r >> 4) * 32 + (r & 3) + 8 * ((r & 15) >> 2) + 4 * (lane >> 5); col = lane & 31; }
            else if (MODE == 1) { row = (r >> 1) * 4 + (lane >> 4); col = (r & 1) * 16 + (lane & 15); }
            else { row = 2 * r + (lane >> 5); col = lane & 31; }
            float* p = tile + row * 128 + col;
            if (MODE == 2) p = hb + (long)qt * 64 * 128 + (wave * 16 + (r >> 1)) * 128 + (r & 1) * 64 + lane;
            if (MODE == 3) __builtin_nontemporal_store(v, p);
            else __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int MODE, bool X>
static void run(const char* name, float* buf, int heads, int T, int nkb) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 grid(heads * nkb);
    hipLaunchKernelGGL((atomic_kernel<MODE, X>), grid, dim3(256), 0, 0, buf, heads, T, nkb, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((atomic_kernel<MODE, X>), grid, dim3(256), 0, 0, buf, heads, T, nkb, 1.0f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    // dwords: per (head, kb): (ntq - kb * ntq / nkb) tiles x 64 x 128
    double dw = 0;
    const int ntq = T / 64;
    for (int kb = 0; kb < nkb; ++kb) dw += (double)(ntq - kb * (ntq / nkb)) * 64 * 128;
    dw *= heads;
    printf("%-34s heads %d T %d key blocks %d: %.3f ms, %.2f G atomics, %.1f G dword/s = %.2f TB/s of operands\n", name, heads, T, nkb, ms, dw / 1e9, dw / ms / 1e6, dw * 4 / ms / 1e9);
}

int main() {
    const int heads = 256, T = 2048;
    float* buf;
    CK(hipMalloc(&buf, (size_t)heads * T * 128 * 4));
    CK(hipMemset(buf, 0, (size_t)heads * T * 128 * 4));
    for (int nkb : {32, 16}) {
        run<0, true>("32x32 layout, XCD-local", buf, heads, T, nkb);
        run<0, false>("32x32 layout, heads fastest", buf, heads, T, nkb);
        run<1, true>("16x16 layout, XCD-local", buf, heads, T, nkb);
        run<2, true>("256 B rows, XCD-local", buf, heads, T, nkb);
        run<3, true>("plain stores 32x32, XCD-local", buf, heads, T, nkb);
    }
    // correctness of the adds: after memset 0 and ONE launch of mode 0 every element of tile qt got (number of key blocks <= qt's block) adds
    CK(hipMemset(buf, 0, (size_t)heads * T * 128 * 4));
    hipLaunchKernelGGL((atomic_kernel<0, true>), dim3(heads * 32), dim3(256), 0, 0, buf, heads, T, 32, 1.0f);
    CK(hipDeviceSynchronize());
    std::vector<float> h((size_t)T * 128);
    CK(hipMemcpy(h.data(), buf + (size_t)5 * T * 128, h.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int q = 0; q < T; ++q)
        for (int d = 0; d < 128; ++d) bad += h[(size_t)q * 128 + d] != (float)(q / 64 + 1);
    printf("sum check (head 5): %d wrong of %d\n", bad, T * 128);
    return 0;
}
