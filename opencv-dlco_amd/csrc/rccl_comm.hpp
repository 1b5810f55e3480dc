// rccl_comm.hpp — direct RCCL collectives (all-gather for a sharded context, sum all-reduce of the F x F partial
// gradient for a replicated one), without a build-time
// dependency: librccl is reached through dlopen (the copy already loaded in the process, e.g.
// torch's, or a path given by the caller).  One communicator per context, created from a
// ncclUniqueId that the host distributes to every rank (MPI, torch.distributed, a file ...).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <string>

namespace dlco {

class RcclComm {
public:
    // 128-byte ncclUniqueId of a new communicator (rank 0 calls this, every rank gets the bytes)
    static void unique_id(void *out128, const char *lib_path);

    RcclComm(const void *id128, int rank, int world, const char *lib_path);
    ~RcclComm();
    // in-place all-gather of buf viewed as [world][bytes_per_rank] on stream s
    void allgather_inplace(void *buf, size_t bytes_per_rank, hipStream_t s);
    // in-place sum all-reduce of `count` floats on stream s (ncclAllReduce, ncclFloat32, ncclSum): every rank ends
    // up with the same bits
    void allreduce_sum_f32(float *buf, size_t count, hipStream_t s);

private:
    void *comm_ = nullptr;
    int rank_ = 0;
};

}  // namespace dlco
