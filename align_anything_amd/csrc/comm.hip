// Collective entry points of the C ABI (aa_comm_init / aa_grad_allreduce_bucket / aa_metrics_allreduce): the
// data-parallel exchange of the DPO / PPO step for hosts that are not Python -- the Python host side uses torch.distributed
// (backend "nccl" = RCCL) for the same three operations (engine.py::GradReducer, trainers/common.py::get_all_reduce_mean).
//
//   reference: DeepSpeed's gradient all-reduce behind `engine.backward / engine.step` (trainers/text_to_text/dpo.py:212-213) and
//   utils/multi_process.py:74-89 (get_all_reduce_mean / _max on the logged scalars).
//
// RCCL is bound at run time (dlopen of librccl.so -- the copy already in the process when the host is a torch process), so
// libaa_hip.so keeps its only link-time dependency, the HIP runtime.  One communicator per library context (csrc/aa_ctx.h; the default
// context = one per process, one process per GPU); the
// collectives are enqueued on the caller's stream, so a bucket's all-reduce on a side stream overlaps with the backward of the
// layers below it exactly as the Python GradReducer does it.
#include "aa_common.h"

#include <dlfcn.h>

namespace {
typedef struct { char internal[128]; } nccl_uid_t;      // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* nccl_comm_t;
enum { kSum = 0, kMax = 2, kAvg = 4, kF32 = 7, kBf16 = 9 };   // ncclRedOp_t / ncclDataType_t values of rccl.h

struct Api {
    void* h = nullptr;
    int (*GetUniqueId)(nccl_uid_t*) = nullptr;
    int (*CommInitRank)(nccl_comm_t*, int, nccl_uid_t, int) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
} g_api;
// the communicator and its rank / world live in the library context (aa_ctx::comm, csrc/aa_ctx.h): one communicator per context

int load_api() {
    if (g_api.h) return AA_OK;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) { aa_set_error("aa_comm: cannot load librccl.so: %s", dlerror()); return AA_ERR_LAUNCH; }
#define AA_SYM(field, name)                                                                  \
    *(void**)(&g_api.field) = dlsym(h, name);                                                \
    if (!g_api.field) { aa_set_error("aa_comm: librccl.so has no %s", name); return AA_ERR_LAUNCH; }
    AA_SYM(GetUniqueId, "ncclGetUniqueId") AA_SYM(CommInitRank, "ncclCommInitRank") AA_SYM(CommDestroy, "ncclCommDestroy")
    AA_SYM(AllReduce, "ncclAllReduce") AA_SYM(Broadcast, "ncclBroadcast") AA_SYM(GetErrorString, "ncclGetErrorString")
#undef AA_SYM
    g_api.h = h;
    return AA_OK;
}
int check(int rc, const char* what) {
    if (rc == 0) return AA_OK;
    aa_set_error("%s: RCCL error %d (%s)", what, rc, g_api.GetErrorString ? g_api.GetErrorString(rc) : "?");
    return AA_ERR_LAUNCH;
}
}  // namespace

// rank 0 creates the 128-byte id and hands it to the other ranks by whatever channel the host has (file, socket, MPI ...)
extern "C" int aa_comm_unique_id(void* id128) {
    if (int rc = load_api()) return rc;
    return check(g_api.GetUniqueId(reinterpret_cast<nccl_uid_t*>(id128)), "aa_comm_unique_id");
}
// one process per GPU: call after hipSetDevice(local_rank)
extern "C" int aa_comm_init(const void* id128, int rank, int world) {
    AA_REQUIRE(world >= 1 && rank >= 0 && rank < world, "aa_comm_init: rank %d of %d", rank, world);
    AA_REQUIRE(aa_ctx_cur()->comm == nullptr, "aa_comm_init: this context already has a communicator (one per context; aa_ctx_create for another)");
    if (int rc = load_api()) return rc;
    nccl_uid_t id;
    memcpy(&id, id128, sizeof(id));
    if (int rc = check(g_api.CommInitRank(&aa_ctx_cur()->comm, world, id, rank), "aa_comm_init")) return rc;
    aa_ctx_cur()->rank = rank; aa_ctx_cur()->world = world;
    return AA_OK;
}
extern "C" int aa_comm_world(int* rank, int* world) {
    if (rank) *rank = aa_ctx_cur()->rank;
    if (world) *world = aa_ctx_cur()->world;
    return AA_OK;
}
void aa_comm_release(aa_ctx* c) {
    if (c->comm) { g_api.CommDestroy(c->comm); c->comm = nullptr; }
    c->rank = 0; c->world = 1;
}
extern "C" int aa_comm_destroy(void) {
    aa_comm_release(aa_ctx_cur());
    return AA_OK;
}
// SUM all-reduce, in place, of one contiguous slice of a flat gradient buffer (dtype 0 = bf16, 1 = fp32) on `stream`: one call per
// decoder layer's ~400 MB bucket as its backward finishes; the 1/world factor is folded into aa_grad_sumsq / aa_adamw_flat (gscale)
extern "C" int aa_grad_allreduce_bucket(void* grads, long count, int dtype, void* stream) {
    AA_REQUIRE(aa_ctx_cur()->comm != nullptr, "aa_grad_allreduce_bucket: call aa_comm_init first");
    AA_REQUIRE(dtype == 0 || dtype == 1, "aa_grad_allreduce_bucket: dtype %d (0 = bf16, 1 = fp32)", dtype);
    if (count <= 0 || aa_ctx_cur()->world == 1) return AA_OK;
    return check(g_api.AllReduce(grads, grads, (size_t)count, dtype == 0 ? kBf16 : kF32, kSum, aa_ctx_cur()->comm, (hipStream_t)stream),
                 "aa_grad_allreduce_bucket");
}
// the step's logged scalars in ONE message: mean (op 0, get_all_reduce_mean) or max (op 1, get_all_reduce_max) over the ranks, fp32, in place
extern "C" int aa_metrics_allreduce(float* vals, int n, int op, void* stream) {
    AA_REQUIRE(aa_ctx_cur()->comm != nullptr, "aa_metrics_allreduce: call aa_comm_init first");
    AA_REQUIRE(op == 0 || op == 1, "aa_metrics_allreduce: op %d (0 = mean, 1 = max)", op);
    if (n <= 0 || aa_ctx_cur()->world == 1) return AA_OK;
    return check(g_api.AllReduce(vals, vals, (size_t)n, kF32, op == 0 ? kAvg : kMax, aa_ctx_cur()->comm, (hipStream_t)stream), "aa_metrics_allreduce");
}
// rank `root`'s buffer to everyone (PPO: reward broadcast when only one rank holds the reward model; initial weight sync)
extern "C" int aa_broadcast(void* buf, long bytes, int root, void* stream) {
    AA_REQUIRE(aa_ctx_cur()->comm != nullptr, "aa_broadcast: call aa_comm_init first");
    if (bytes <= 0 || aa_ctx_cur()->world == 1) return AA_OK;
    return check(g_api.Broadcast(buf, buf, (size_t)bytes, 0 /* ncclInt8 */, root, aa_ctx_cur()->comm, (hipStream_t)stream), "aa_broadcast");
}
