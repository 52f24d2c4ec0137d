// Thin RCCL wrappers of the C-ABI (`svae_allreduce_*` taking an ncclComm_t) for hosts that drive the
// data-parallel exchange without torch.distributed: communicator set-up from a 128-byte unique id, the in-place sum
// all-reduce of a gradient bucket, and the reduce-scatter / all-gather pair of the sharded decoder.linear2 bucket.
// The reference is single-process (no counterpart); the Python side of this build uses torch.distributed's RCCL
// backend for the same collectives (svae_cf_amd/dist.py).
//
// librccl is loaded on first use (dlopen), not linked: the library stays loadable on a host without RCCL, and a
// process that already carries torch's RCCL does not get a second copy mapped at start-up.
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "common.h"

namespace {

constexpr int ID_BYTES = 128;                 // NCCL_UNIQUE_ID_BYTES
struct UniqueId { char internal[ID_BYTES]; };
typedef void* comm_t;
enum { kFloat32 = 7, kSum = 0 };              // ncclFloat32, ncclSum (rccl.h)

struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(comm_t*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};
Rccl g_rccl;
std::once_flag g_once;

char g_load_error[512] = "(unknown)";   // why the load failed: dlerror() clears itself when read, so it is kept here

void load_rccl() {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    size_t used = 0;
    for (const char* n : names) {
        g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.h) break;
        const char* e = dlerror();
        int w = snprintf(g_load_error + used, sizeof(g_load_error) - used, "%s%s: %s", used ? "; " : "", n,
                         e ? e : "(unknown)");
        if (w > 0) used = used + (size_t)w < sizeof(g_load_error) ? used + (size_t)w : sizeof(g_load_error) - 1;
    }
    if (!g_rccl.h) return;
    auto sym = [&](const char* s) { return dlsym(g_rccl.h, s); };
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
    g_rccl.ReduceScatter = (decltype(g_rccl.ReduceScatter))sym("ncclReduceScatter");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))sym("ncclAllGather");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce &&
                g_rccl.ReduceScatter && g_rccl.AllGather && g_rccl.GetErrorString;
}

int need_rccl(const char* who) {
    std::call_once(g_once, load_rccl);
    if (!g_rccl.ok) return svae_fail(-3, "%s: librccl could not be loaded (%s)", who, g_rccl.h ? "symbols missing" : g_load_error);
    return 0;
}
int rc_of(int r, const char* who) {
    return r == 0 ? 0 : svae_fail(1000 + r, "%s: RCCL: %s", who, g_rccl.GetErrorString(r));
}

}  // namespace

// 128 bytes that rank 0 hands to every rank (over whatever channel the host has) before svae_rccl_comm_create.
extern "C" int svae_rccl_unique_id(void* out128) {
    SV_CHECK_ARG(out128, "svae_rccl_unique_id: null pointer");
    if (int e = need_rccl("svae_rccl_unique_id")) return e;
    return rc_of(g_rccl.GetUniqueId((UniqueId*)out128), "svae_rccl_unique_id");
}
// One communicator per process and GPU (the device current at the call); collective over all nranks callers.
extern "C" int svae_rccl_comm_create(void** comm, int nranks, int rank, const void* id128) {
    SV_CHECK_ARG(comm && id128 && nranks >= 1 && rank >= 0 && rank < nranks, "svae_rccl_comm_create: bad arguments");
    if (int e = need_rccl("svae_rccl_comm_create")) return e;
    UniqueId id;
    memcpy(&id, id128, ID_BYTES);
    return rc_of(g_rccl.CommInitRank((comm_t*)comm, nranks, id, rank), "svae_rccl_comm_create");
}
extern "C" int svae_rccl_comm_destroy(void* comm) {
    SV_CHECK_ARG(comm, "svae_rccl_comm_destroy: null pointer");
    if (int e = need_rccl("svae_rccl_comm_destroy")) return e;
    return rc_of(g_rccl.CommDestroy((comm_t)comm), "svae_rccl_comm_destroy");
}
// In-place sum all-reduce of count floats (a gradient bucket of the flat buffer), enqueued on `stream`.
extern "C" int svae_allreduce_sum_f32(void* comm, float* buf, long long count, void* stream) {
    SV_CHECK_ARG(comm && buf && count > 0, "svae_allreduce_sum_f32: bad arguments");
    if (int e = need_rccl("svae_allreduce_sum_f32")) return e;
    return rc_of(g_rccl.AllReduce(buf, buf, (size_t)count, kFloat32, kSum, (comm_t)comm, (hipStream_t)stream),
                 "svae_allreduce_sum_f32");
}
// The sharded bucket (decoder.linear2): bucket = nranks shards of shard_count floats; rank r's shard of the sum
// lands in place (bucket + r * shard_count), the rest of the bucket is left with partial sums ...
extern "C" int svae_reduce_scatter_sum_f32(void* comm, float* bucket, long long shard_count, int rank, void* stream) {
    SV_CHECK_ARG(comm && bucket && shard_count > 0 && rank >= 0, "svae_reduce_scatter_sum_f32: bad arguments");
    if (int e = need_rccl("svae_reduce_scatter_sum_f32")) return e;
    return rc_of(g_rccl.ReduceScatter(bucket, bucket + (size_t)rank * shard_count, (size_t)shard_count, kFloat32, kSum,
                                      (comm_t)comm, (hipStream_t)stream), "svae_reduce_scatter_sum_f32");
}
// ... and after the optimizer step over that shard, every rank's shard of the parameters goes to every rank, in place.
extern "C" int svae_allgather_f32(void* comm, float* bucket, long long shard_count, int rank, void* stream) {
    SV_CHECK_ARG(comm && bucket && shard_count > 0 && rank >= 0, "svae_allgather_f32: bad arguments");
    if (int e = need_rccl("svae_allgather_f32")) return e;
    return rc_of(g_rccl.AllGather(bucket + (size_t)rank * shard_count, bucket, (size_t)shard_count, kFloat32,
                                  (comm_t)comm, (hipStream_t)stream), "svae_allgather_f32");
}
