// Data-parallel gradient exchange: ms_comm_* / ms_allreduce_f32 over RCCL (include/msynth.h).
//
// RCCL is bound at run time from the instance already loaded into the process (torch's "nccl" backend is
// RCCL on ROCm and ships its own copy next to its own HIP runtime: linking a second copy at build time
// would put two RCCLs / two HIP runtimes into one process).  Only the five entry points the gradient
// exchange needs are resolved.  One communicator per process, one process per GPU; the all-reduce is a
// plain stream-ordered ncclAllReduce(float, sum), in place, on the caller's stream.
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include <mutex>

#include "ms_common.h"

namespace {

// the RCCL C API subset used here (rccl.h: ncclUniqueId is 128 opaque bytes, ncclFloat32 = 7, ncclSum = 0)
struct UniqueId { char internal[MS_COMM_ID_BYTES]; };
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(void**, int, UniqueId, int);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*CommDestroyFn)(void*);
typedef const char* (*GetErrorStringFn)(int);
constexpr int kFloat32 = 7, kSum = 0;

struct Api {
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    AllReduceFn all_reduce = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    GetErrorStringFn get_error_string = nullptr;
    bool ok = false;
};

Api g_api;
std::once_flag g_once;
char g_err[512] = "";

void set_err(const char* what, int rc) {
    const char* txt = (g_api.get_error_string && rc != 0) ? g_api.get_error_string(rc) : "";
    snprintf(g_err, sizeof(g_err), "%s%s%s", what, txt[0] ? ": " : "", txt);
}

void bind() {
    void* h = nullptr;
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names)                       // an instance the process already holds (torch's)
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char* n : names)                       // else from the loader path
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!h) { set_err("librccl.so.1 not found", 0); return; }
    g_api.get_unique_id = reinterpret_cast<GetUniqueIdFn>(dlsym(h, "ncclGetUniqueId"));
    g_api.comm_init_rank = reinterpret_cast<CommInitRankFn>(dlsym(h, "ncclCommInitRank"));
    g_api.all_reduce = reinterpret_cast<AllReduceFn>(dlsym(h, "ncclAllReduce"));
    g_api.comm_destroy = reinterpret_cast<CommDestroyFn>(dlsym(h, "ncclCommDestroy"));
    g_api.get_error_string = reinterpret_cast<GetErrorStringFn>(dlsym(h, "ncclGetErrorString"));
    g_api.ok = g_api.get_unique_id && g_api.comm_init_rank && g_api.all_reduce && g_api.comm_destroy;
    if (!g_api.ok) set_err("RCCL entry points missing", 0);
}

bool api() {
    std::call_once(g_once, bind);
    return g_api.ok;
}

}  // namespace

struct ms_comm {
    void* nccl;
    int world, rank;
};

extern "C" {

const char* ms_comm_last_error(void) { return g_err; }

int ms_comm_unique_id(void* id_out) {
    if (!id_out) return MS_ERR_INVALID_ARG;
    if (!api()) return MS_ERR_COMM;
    UniqueId id;
    const int rc = g_api.get_unique_id(&id);
    if (rc != 0) { set_err("ncclGetUniqueId", rc); return MS_ERR_COMM; }
    memcpy(id_out, &id, MS_COMM_ID_BYTES);
    return MS_OK;
}

int ms_comm_init(const void* id, int32_t world, int32_t rank, ms_comm_t* out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return MS_ERR_INVALID_ARG;
    if (!api()) return MS_ERR_COMM;
    UniqueId uid;
    memcpy(&uid, id, MS_COMM_ID_BYTES);
    void* c = nullptr;
    const int rc = g_api.comm_init_rank(&c, world, uid, rank);
    if (rc != 0 || !c) { set_err("ncclCommInitRank", rc); return MS_ERR_COMM; }
    *out = new ms_comm{c, world, rank};
    return MS_OK;
}

int ms_comm_world(ms_comm_t comm) { return comm ? comm->world : MS_ERR_INVALID_ARG; }
int ms_comm_rank(ms_comm_t comm) { return comm ? comm->rank : MS_ERR_INVALID_ARG; }

int ms_allreduce_f32(ms_comm_t comm, float* buf, int64_t n, ms_stream_t stream) {
    if (!comm || !buf || n < 0) return MS_ERR_INVALID_ARG;
    if (n == 0) return MS_OK;
    const int rc = g_api.all_reduce(buf, buf, (size_t)n, kFloat32, kSum, comm->nccl,
                                    reinterpret_cast<hipStream_t>(stream));
    if (rc != 0) { set_err("ncclAllReduce", rc); return MS_ERR_COMM; }
    return MS_OK;
}

int ms_comm_destroy(ms_comm_t comm) {
    if (!comm) return MS_ERR_INVALID_ARG;
    const int rc = g_api.comm_destroy(comm->nccl);
    delete comm;
    if (rc != 0) { set_err("ncclCommDestroy", rc); return MS_ERR_COMM; }
    return MS_OK;
}

}  // extern "C"
