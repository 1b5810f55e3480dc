// rccl_comm.cpp — see rccl_comm.hpp.
#include "rccl_comm.hpp"

#include "dlco_internal.hpp"

#include <dlfcn.h>

#include <cstring>

namespace dlco {

namespace {

struct UniqueId { char internal[128]; };                      // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
using comm_t = void *;

struct Api {
    void *h = nullptr;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(comm_t *, int, UniqueId, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, comm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

Api &api(const char *lib_path)
{
    static Api a;
    if (a.h) return a;
    void *h = nullptr;
    // prefer the copy that is already in the process (torch loads its own librccl)
    for (const char *name : {"librccl.so.1", "librccl.so"}) {
        h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
        if (h) break;
    }
    if (!h && lib_path && *lib_path) h = dlopen(lib_path, RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
    }
    if (!h) throw Error(-6, "librccl could not be loaded (give its path, or use the all-gather callback)");
    auto sym = [&](const char *n) {
        void *p = dlsym(h, n);
        if (!p) throw Error(-6, std::string("librccl lacks ") + n);
        return p;
    };
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(sym("ncclAllGather"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
    a.h = h;
    return a;
}

void check(Api &a, int rc, const char *what)
{
    if (rc != 0) throw Error(-6, std::string(what) + ": " + (a.GetErrorString ? a.GetErrorString(rc) : "RCCL error"));
}

}  // namespace

void RcclComm::unique_id(void *out128, const char *lib_path)
{
    Api &a = api(lib_path);
    UniqueId id;
    std::memset(&id, 0, sizeof(id));
    check(a, a.GetUniqueId(&id), "ncclGetUniqueId");
    std::memcpy(out128, &id, sizeof(id));
}

RcclComm::RcclComm(const void *id128, int rank, int world, const char *lib_path) : rank_(rank)
{
    Api &a = api(lib_path);
    UniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    comm_t c = nullptr;
    check(a, a.CommInitRank(&c, world, id, rank), "ncclCommInitRank");
    comm_ = c;
}

RcclComm::~RcclComm()
{
    if (comm_) (void)api(nullptr).CommDestroy(comm_);
}

void RcclComm::allgather_inplace(void *buf, size_t bytes_per_rank, hipStream_t s)
{
    Api &a = api(nullptr);
    // bytes are moved as ncclInt8 (= 0): the library never interprets the payload
    check(a, a.AllGather(static_cast<char *>(buf) + (size_t)rank_ * bytes_per_rank, buf, bytes_per_rank, 0, comm_, s),
          "ncclAllGather");
}

void RcclComm::allreduce_sum_f32(float *buf, size_t count, hipStream_t s)
{
    Api &a = api(nullptr);
    constexpr int kFloat32 = 7, kSum = 0;                      // ncclFloat32, ncclSum (nccl.h)
    check(a, a.AllReduce(buf, buf, count, kFloat32, kSum, comm_, s), "ncclAllReduce");
}

}  // namespace dlco
