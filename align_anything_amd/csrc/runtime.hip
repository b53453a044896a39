// Library runtime: error reporting, device info, HIP-event timing on an explicit stream, and two
// hardware probes (MFMA fragment layout, ds_read_b64_tr_b16 lane mapping) that the GPU test-suite
// uses to pin the layout assumptions the GEMM/attention kernels are built on.
#include "aa_common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void aa_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* aa_last_error(void) { return g_err; }
extern "C" int aa_version(void) { return 101; }

// ---- library contexts (csrc/aa_ctx.h): the state behind the set_* switches, plan records and the communicator
static aa_ctx g_default_ctx;
static thread_local aa_ctx* t_ctx = nullptr;
aa_ctx* aa_ctx_cur() { return t_ctx ? t_ctx : &g_default_ctx; }

extern "C" int aa_ctx_create(void** ctx) {
    AA_REQUIRE(ctx != nullptr, "aa_ctx_create: ctx is null");
    *ctx = new aa_ctx();
    return AA_OK;
}
// NULL = back to the process-wide default context.  Per thread, like hipSetDevice.
extern "C" int aa_ctx_set_current(void* ctx) { t_ctx = static_cast<aa_ctx*>(ctx); return AA_OK; }
extern "C" int aa_ctx_get_current(void** ctx) {
    AA_REQUIRE(ctx != nullptr, "aa_ctx_get_current: ctx is null");
    *ctx = t_ctx;            // NULL while the thread uses the default context
    return AA_OK;
}
extern "C" int aa_ctx_destroy(void* ctx) {
    aa_ctx* c = static_cast<aa_ctx*>(ctx);
    AA_REQUIRE(c != nullptr && c != &g_default_ctx, "aa_ctx_destroy: not a context made by aa_ctx_create");
    if (t_ctx == c) t_ctx = nullptr;
    aa_comm_release(c);
    delete c;
    return AA_OK;
}

extern "C" int aa_device_info(int* cu_count, int* lds_per_cu, int* wave_size, char* arch, int arch_len) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) { aa_set_error("aa_device_info: %s", hipGetErrorString(e)); return AA_ERR_LAUNCH; }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) { aa_set_error("aa_device_info: %s", hipGetErrorString(e)); return AA_ERR_LAUNCH; }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (lds_per_cu) *lds_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
    if (wave_size) *wave_size = prop.warpSize;
    if (arch && arch_len > 0) { strncpy(arch, prop.gcnArchName, arch_len - 1); arch[arch_len - 1] = 0; }
    return AA_OK;
}

// ---- HIP events on the caller's stream (torch.cuda.Event only sees torch's current stream)
extern "C" int aa_event_create(void** ev) {
    hipEvent_t e;
    hipError_t r = hipEventCreate(&e);
    if (r != hipSuccess) { aa_set_error("aa_event_create: %s", hipGetErrorString(r)); return AA_ERR_LAUNCH; }
    *ev = (void*)e;
    return AA_OK;
}
extern "C" int aa_event_record(void* ev, void* stream) {
    hipError_t r = hipEventRecord((hipEvent_t)ev, (hipStream_t)stream);
    if (r != hipSuccess) { aa_set_error("aa_event_record: %s", hipGetErrorString(r)); return AA_ERR_LAUNCH; }
    return AA_OK;
}
extern "C" int aa_event_elapsed_ms(void* a, void* b, float* ms) {
    hipError_t r = hipEventSynchronize((hipEvent_t)b);
    if (r == hipSuccess) r = hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b);
    if (r != hipSuccess) { aa_set_error("aa_event_elapsed_ms: %s", hipGetErrorString(r)); return AA_ERR_LAUNCH; }
    return AA_OK;
}
extern "C" int aa_event_destroy(void* ev) { hipEventDestroy((hipEvent_t)ev); return AA_OK; }

// ---- probe 1: which (row, col) of D does (lane, reg) hold for v_mfma_f32_16x16x32_bf16 ?
// A[i][k] = (i == row_sel && k == 0), B[k][j] = (k == 0) * (j + 1)  ->  D[row_sel][j] = j + 1.
// Each lane writes its 4 accumulator values; the host decodes the map.
__global__ void probe_mfma_kernel(float* out, int row_sel) {
    const int lane = threadIdx.x;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)0.f; b[e] = (__bf16)0.f; }
    // assumed operand map: lane holds A[i = lane&15][k = (lane>>4)*8 + e], B[k = (lane>>4)*8 + e][j = lane&15]
    if ((lane >> 4) == 0) {
        if ((lane & 15) == row_sel) a[0] = (__bf16)1.f;
        b[0] = (__bf16)(float)((lane & 15) + 1);
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}
extern "C" int aa_probe_mfma(float* out256, int row_sel, void* stream) {
    hipLaunchKernelGGL(probe_mfma_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out256, row_sel);
    AA_CHECK_LAUNCH("aa_probe_mfma");
    return AA_OK;
}

// ---- probe 2: ds_read_b64_tr_b16.  LDS is filled with lds[i] = i (bf16-exact for i < 256);
// lane l reads from byte address addr[l]; out[l*4 + j] = element j it received.
__global__ void probe_tr_kernel(const int* addr, float* out) {
    __shared__ __attribute__((aligned(16))) bf16_t lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = f2bf((float)(i & 255));
    __syncthreads();
    const char* base = reinterpret_cast<const char*>(lds);
    bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
        (__attribute__((address_space(3))) bf16x4*)(base + addr[threadIdx.x]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (float)v[j];
}
extern "C" int aa_probe_tr16(const int* addr64, float* out256, void* stream) {
    hipLaunchKernelGGL(probe_tr_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, addr64, out256);
    AA_CHECK_LAUNCH("aa_probe_tr16");
    return AA_OK;
}

// ---- a one-GPU MODEL of what a resident collective costs the compute stream (tools/dp_shadow.py).
// RCCL's kernels hold C compute units for the length of the backward pass and move the gradient buckets through HBM at link speed.  Two pieces
// reproduce that on one device: (1) a stream whose kernels may only use a subset of the CUs (hipExtStreamCreateWithCUMask) -- the compute stream with
// the collective's CUs taken away; (2) a traffic kernel of C long-lived workgroups that stream `bytes` from src to dst `passes` times -- the
// collective's HBM reads / writes, at the ~50 GB/s one CU sustains, from CUs the GEMMs then cannot use.
extern "C" int aa_stream_create_cu_mask(const unsigned int* mask, int words, void** stream_out) {
    AA_REQUIRE(mask != nullptr && words > 0 && stream_out != nullptr, "aa_stream_create_cu_mask: mask / words / stream_out");
    hipStream_t st;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask);
    if (e != hipSuccess) { aa_set_error("aa_stream_create_cu_mask: %s", hipGetErrorString(e)); return AA_ERR_LAUNCH; }
    *stream_out = (void*)st;
    return AA_OK;
}
extern "C" int aa_stream_destroy(void* stream) {
    hipError_t e = hipStreamDestroy((hipStream_t)stream);
    if (e != hipSuccess) { aa_set_error("aa_stream_destroy: %s", hipGetErrorString(e)); return AA_ERR_LAUNCH; }
    return AA_OK;
}
__global__ __launch_bounds__(256) void shadow_traffic_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long n16, int passes) {
    for (int p = 0; p < passes; ++p)
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
            f32x4 v = __builtin_nontemporal_load(src + i);
            v[0] += (float)p;                          // passes must not collapse into one
            __builtin_nontemporal_store(v, dst + i);
        }
}
extern "C" int aa_shadow_traffic(const void* src, void* dst, long bytes, int workgroups, int passes, void* stream) {
    AA_REQUIRE(src && dst && bytes >= 16 && (bytes & 15) == 0 && workgroups > 0 && passes > 0, "aa_shadow_traffic: 16-byte multiples, workgroups, passes > 0");
    hipLaunchKernelGGL(shadow_traffic_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, (const f32x4*)src, (f32x4*)dst, bytes / 16, passes);
    AA_CHECK_LAUNCH("aa_shadow_traffic");
    return AA_OK;
}
