// Library context (SURVEY.md section 8(b) proposed `aa_ctx*`): every piece of state the C ABI used to keep in
// process globals -- GEMM tile / raster / fusion switches, the per-shape SwiGLU-backward plan records, the attention implementation
// switch, the optimizer-kernel switch, and the RCCL communicator with its rank / world -- lives in one `aa_ctx`.
//
// The entry points keep their signatures: they act on the calling THREAD's current context (`aa_ctx_set_current`, like hipSetDevice /
// hipCtxSetCurrent), and a thread that never set one uses the process-wide default context, which is exactly the old behaviour.  A host
// that drives two devices, or two models that want different plans, from one process creates one context per (thread, device / model):
//     aa_ctx* c; aa_ctx_create(&c); aa_ctx_set_current(c); hipSetDevice(1); aa_comm_init(id, rank, world); aa_gemm_set_group(3); ...
// Contexts are not shared between threads concurrently (no internal locking): one host thread per context at a time, the model the
// reference has anyway (one process per GPU, a single host thread driving its streams).
#pragma once

struct AaGluPlan { int mb, F, K, fused; unsigned long stamp; };
constexpr int AA_GLU_PLANS = 64;

struct aa_ctx {
    // csrc/gemm.hip
    int gm = 0;                       // tile-group height of the grouped tile order; 0 = heuristic
    int force_tile = -2;              // -2: read AA_GEMM_TILE once; -1: heuristic
    int fuse = -1;                    // -1: read AA_GEMM_FUSE once
    int glu_mode = -1;                // AA_GLU_BWD: -1 follow the records, 0 unfused, 1 fused
    bool glu_mode_read = false;
    AaGluPlan glu_plans[AA_GLU_PLANS];
    int glu_nplans = 0;
    unsigned long glu_clock = 0;
    // csrc/attention.hip
    int attn_impl = -1;               // -1: read AA_ATTN128 once
    // csrc/optim.hip
    int adam_thin = 0;
    // csrc/comm.hip
    void* comm = nullptr;             // ncclComm_t
    int rank = 0, world = 1;
};

aa_ctx* aa_ctx_cur();                 // runtime.hip: the calling thread's current context, or the process default
void aa_comm_release(aa_ctx* c);      // comm.hip: destroy c's communicator, if any
