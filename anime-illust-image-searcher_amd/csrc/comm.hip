// comm.hip -- the one collective of the indexing path behind the C ABI: an all-gather of fixed-width tag rows (or feature rows) in rank
// order == file order (the loop of tagging.py:276-359 cut into contiguous blocks, one process per GPU).  The Python
// CLIs reach RCCL through torch.distributed; a host in another language binds these entry points instead.  RCCL is resolved at run time:
// from the process image first (a PyTorch process has its own librccl mapped -- a second copy must not be brought in), then librccl.so.1.
#include <dlfcn.h>

#include <rccl/rccl.h>

#include "common.h"

#include "../../include/hip_tagsearch.h"

namespace hipts {
namespace {

struct Rccl {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

template <class F>
bool resolve(void* lib, const char* name, F& fn) {
    void* p = dlsym(RTLD_DEFAULT, name);
    if (!p && lib) p = dlsym(lib, name);
    fn = reinterpret_cast<F>(p);
    return p != nullptr;
}

Rccl& rccl() {
    static Rccl* r = [] {
        Rccl* x = new Rccl;
        void* lib = nullptr;
        if (!dlsym(RTLD_DEFAULT, "ncclAllGather")) {
            lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        }
        x->ok = resolve(lib, "ncclGetUniqueId", x->GetUniqueId) & resolve(lib, "ncclCommInitRank", x->CommInitRank) &
                resolve(lib, "ncclCommDestroy", x->CommDestroy) & resolve(lib, "ncclAllGather", x->AllGather) &
                resolve(lib, "ncclGetErrorString", x->GetErrorString);
        return x;
    }();
    return *r;
}

#define HIPTS_RCCL(expr)                                                                                              \
    do {                                                                                                              \
        const ncclResult_t r_ = (expr);                                                                               \
        if (r_ != ncclSuccess) return set_error(HIPTS_ERR_HIP, "%s failed: %s", #expr, rccl().GetErrorString(r_));    \
    } while (0)

}  // namespace
}  // namespace hipts

using namespace hipts;

struct hipts_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

extern "C" int hipts_comm_unique_id(uint8_t* id_out, size_t bytes) {
    HIPTS_REQUIRE(id_out && bytes >= sizeof(ncclUniqueId), "hipts_comm_unique_id: the buffer must hold %zu bytes", sizeof(ncclUniqueId));
    HIPTS_REQUIRE(rccl().ok, "RCCL (librccl.so.1) is not available in this process");
    ncclUniqueId id;
    HIPTS_RCCL(rccl().GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return HIPTS_OK;
}

extern "C" int hipts_comm_create(const uint8_t* id, size_t bytes, int rank, int world, int device, hipts_comm_t** out) {
    HIPTS_REQUIRE(id && out && bytes >= sizeof(ncclUniqueId) && world >= 1 && rank >= 0 && rank < world, "hipts_comm_create: bad arguments");
    HIPTS_REQUIRE(rccl().ok, "RCCL (librccl.so.1) is not available in this process");
    HIPTS_TRY(use_device(device));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    hipts_comm* c = new hipts_comm;
    c->rank = rank; c->world = world; c->device = device;
    const ncclResult_t r = rccl().CommInitRank(&c->comm, world, uid, rank);
    if (r != ncclSuccess) {
        delete c;
        return set_error(HIPTS_ERR_HIP, "ncclCommInitRank failed: %s", rccl().GetErrorString(r));
    }
    *out = c;
    return HIPTS_OK;
}

extern "C" int hipts_comm_destroy(hipts_comm_t* c) {
    if (!c) return HIPTS_OK;
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    delete c;
    return HIPTS_OK;
}

extern "C" int hipts_allgather_rows(hipts_comm_t* c, const int32_t* rows_device, int64_t rows_per_rank, int row_width, int32_t* out_device,
                                    void* stream) {
    HIPTS_REQUIRE(c && c->comm && rows_device && out_device && rows_per_rank >= 0 && row_width >= 1, "hipts_allgather_rows: bad arguments");
    HIPTS_TRY(use_device(c->device));
    if (rows_per_rank == 0) return HIPTS_OK;
    HIPTS_RCCL(rccl().AllGather(rows_device, out_device, (size_t)rows_per_rank * row_width, ncclInt32, c->comm, (hipStream_t)stream));
    return HIPTS_OK;
}
